for t in 512 256 384 768; do echo "target $t"; BIST_GEMM_SPLIT_TARGET=$t BIG=1 python scripts/bench_gemm.py 2>&1 | grep "dW"; done
