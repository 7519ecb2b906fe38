#!/bin/bash
# tools/sweep_engines.sh — v_dot4 vs MFMA pass-0 engine over every fused configuration (GPU box)
for P in 1 2 3 4 5 6; do for F in "" "--fir9"; do for A in std fast; do
  python tools/ab_engines.py --passes $P $F --atan $A --rounds 12 --burst 4 2>/dev/null | grep "^P="
done; done; done
