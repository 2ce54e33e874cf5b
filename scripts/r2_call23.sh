timeout 900 python -X faulthandler -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "parallel_resolve or cpp_mirror or random_programs" 2>&1 | tail -15 | cut -c1-400
