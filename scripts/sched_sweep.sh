#!/bin/bash
# panel-schedule sweep at N=16384 (one process per setting; ~1 s each)
for cfg in "nb_auto=0" "nb_auto=1" "nb_auto=1 nb_big_until=4096" "nb_auto=1 nb_big_until=12288" "nb_auto=1 nb_small_from=4096" "nb_auto=1 nb_small_from=1024" "nb_auto=1 nb_big=768" "nb_auto=1 nb_big=1024 nb_big_until=10240"; do
  echo "== $cfg"; timeout -k 10 100 python scripts/prof_target.py 16384 8 1024 3 $cfg | sed -n '1p;3p;5p;6p;7p'
done
