cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/fuzz_parity.py --seconds 100 --seed 303 2>&1 | tail -1
