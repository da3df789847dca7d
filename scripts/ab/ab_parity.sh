# A/B of library variants on one box + a quick parity run of the LAST variant: scripts/ab/ab_parity.sh A B [A B ...]
bash scripts/ab/kernel_ab2.sh "$@"
last="${@: -1}"
MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$last.so timeout 900 python -m pytest tests/test_gpu_operator_parity.py tests/test_gpu_bruteforce.py tests/test_gpu_known_answers.py -m gpu -x -q 2>&1 | tail -2
