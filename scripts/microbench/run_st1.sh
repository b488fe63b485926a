for b in 16 64; do for dbg in 0 1 2 4 3 7; do echo "B=$b dbg=$dbg"; B=$b BIST_ST1_DBG=$dbg python scripts/bench_st1.py 2>&1 | grep "st1 fwd"; done; done
