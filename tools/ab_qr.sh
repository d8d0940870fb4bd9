for rep in 1 2; do for f in tools/ab/*.so; do echo -n "$(basename $f) : "; DQMC_HIP_LIB="$PWD/$f" timeout -k 10 100 python tools/time_qr.py 2>/dev/null | tail -1; done; done
