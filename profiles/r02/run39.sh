# order replay with reusable bucket arrays, pinned staging, hashing inside the group threads
python profiles/stream_probe.py 2>&1 | grep -E "per frame|frames"
python -m pytest tests/test_pipeline.py tests/test_kitti_io.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
