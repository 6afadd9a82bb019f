TUNE_ARGS="--what 14 --runs 5" bash scripts/run_variants.sh bit0 new bit0 new
python3 scripts/tune_elist.py --lib variants/bit0.so --runs 5 --tag "elist bit0" 2>&1 | grep -v amdgpu
python3 scripts/tune_elist.py --lib variants/new.so --runs 5 --tag "elist new" 2>&1 | grep -v amdgpu
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
