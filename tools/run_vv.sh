for b in tools/bin/vv_*; do $b 8; done
