cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_segments.py -x -q 2>&1 | tail -15
python -m pytest tests/test_gpu_batch.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -5
