cd $GRAFT_REPO_ROOT
python -m pytest tests/test_golden_fixtures.py -x -q -m gpu -k "ac_256_HPR" 2>&1 | tail -30
ZEN_HIP_OPTIONS="no_median_bits=1" python -m pytest tests/test_golden_fixtures.py -x -q -m gpu -k "ac_256" 2>&1 | tail -3
