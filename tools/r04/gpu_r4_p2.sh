#!/bin/bash
# key-point detector with one image channel + smoke
python -m pytest tests/test_kp_detector.py -q -x -m gpu -s 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
