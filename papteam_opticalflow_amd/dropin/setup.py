"""Builds the drop-in `pyflow` extension in place (host glue only; links libpapof.so by rpath).
    python setup.py build_ext --inplace
"""
import os

import numpy
from Cython.Build import cythonize
from setuptools import Extension, setup

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")

ext = Extension(
    "pyflow",
    [os.path.join(HERE, "pyflow.pyx")],
    include_dirs=[numpy.get_include(), os.path.join(ROOT, "include")],
    library_dirs=[CSRC],
    libraries=["papof"],
    runtime_library_dirs=["$ORIGIN/../csrc"],
    extra_compile_args=["-O2"],
    define_macros=[("NPY_NO_DEPRECATED_API", "NPY_1_7_API_VERSION")],
)

setup(name="pyflow", version="0.1.0", ext_modules=cythonize([ext], language_level=3, quiet=True),
      script_args=["build_ext", "--inplace", "--build-temp", os.path.join(HERE, "build")] if __name__ == "__main__" and len(os.sys.argv) == 1 else None)
