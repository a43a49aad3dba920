for R in 64 96 128 192 256; do echo "R $R"; RTGPU_LL_R=$R timeout -k 10 100 python -u tools/opt_ab.py temple 3840 2160 "" 2>&1 | tail -1 || exit 1; done
