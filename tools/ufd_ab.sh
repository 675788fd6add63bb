for m in direct tiled sr8; do echo "== $m"; HAVATAR_UFD_DOWN2=$m timeout 300 python tools/bench_ops.py 2>&1 | grep -E "upfirdn2d (down2 k4|blur k4 \[64|up2 k4)"; done
