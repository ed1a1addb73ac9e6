cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 tools/batch_probe.py 240 15 16,32 | tail -2; done
python3 tools/batch_probe.py 240 8 32 | tail -1
python3 tools/collection_probe.py 240 100 16 2>&1 | tail -6
python3 tools/collection_probe.py 480 64 16 2>&1 | tail -6
