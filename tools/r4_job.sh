timeout 900 python -m pytest tests/test_compositing.py tests/test_fog.py tests/test_subsurface.py -m gpu -x -q 2>&1 | tail -3
