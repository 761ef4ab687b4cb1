#!/bin/bash
timeout 1200 python -m pytest tests -m gpu -x -q -k "center or estimate or full_size or dropin or auto_interp" 2>&1 | tail -3
bash tools/r3_est_prof.sh r3c_center 2>&1 | grep "^{\|k_me_sum_fin" | cut -c1-300
