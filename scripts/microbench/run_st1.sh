for t in 128 32; do for gc in 0 2 1; do echo "T=$t gc=$gc"; B=16 T=$t BIST_ST1_GC=$gc python scripts/bench_st1.py 2>&1 | grep "st1 fwd"; done; done
