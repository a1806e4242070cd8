cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "mask_bits or offline or golden or config" 2>&1 | tail -3
bash tools/prof_offline.sh | cut -c1-150 | grep -i "transpose\|istft_kernel<14\|Name"
