#!/bin/bash
exec < /dev/null
cd /root/repo
O=/root/repo/gpurun_out/r03
mkdir -p $O
for w in 1 2 3 4; do
  ( timeout 300 python -X faulthandler -m pytest tests/test_coupled.py -m gpu -q --timeout 120 -x -k "hard_spread_random or soft_and_hard" > $O/dbg_w$w.txt 2>&1; echo "rc=$?" >> $O/dbg_w$w.txt ) &
done
wait
for w in 1 2 3 4; do echo "== worker $w"; tail -25 $O/dbg_w$w.txt; done
