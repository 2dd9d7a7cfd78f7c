#!/bin/bash
# panel-schedule sweep at N=16384, fused pipeline (one process per setting; A/B must be within one box)
for cfg in "nb_auto=1" "nb_small_from=4096" "nb_small_from=6144" "nb_small_from=8192" "nb_big_until=6144" "nb_big_until=10240" "nb_big_until=6144 nb_small_from=4096" "nb_big=1024" "nb_big=768" "nb_auto=1"; do
  echo "== $cfg: $(timeout -k 10 100 python scripts/prof_target.py 16384 8 1024 8 $cfg | sed -n '1p' | cut -c28-72)"
done
