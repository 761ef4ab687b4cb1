#!/bin/bash
timeout 1200 python -m pytest tests -m gpu -x -q -k "center or estimate or hist or full_size or dropin or auto_interp" 2>&1 | tail -3
bash tools/r3_est_prof.sh r3e2 2>&1 | grep -v "^E2026\|^W2026" | grep "k_me_hist\|k_me_first\|k_me_leaves\|k_me_compact\|k_me_sum_fin\|^{" | cut -c1-700
