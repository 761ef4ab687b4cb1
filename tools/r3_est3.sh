#!/bin/bash
timeout 1200 python -m pytest tests -m gpu -x -q -k "plateau or center or estimate or hist or full_size or dropin or auto_interp" 2>&1 | tail -15
python tools/est_probe.py --no-psk 2>/dev/null | grep '^{' | cut -c1-700
