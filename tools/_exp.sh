export CRT_WATCHDOG_SECONDS=5
L=cudaraytracing_amd/libcrt_amd.so
cp build_variants/libB.so $L
timeout -k 10 150 python -u -c "import __graft_entry__ as g; g.smoke()" || exit 1
for round in 1 2; do
cp build_variants/libA.so $L; echo "== A (HEAD)"; timeout -k 10 300 python tools/knob_sweep.py 16 "" || exit 1
cp build_variants/libB.so $L; echo "== B"; timeout -k 10 600 python tools/knob_sweep.py 16 "" "DIRECT_POST=0" "SWING_FILL=192" "TRAV_WAVES=14,SWING_WAVES=7" "TRAV_WAVES=12,SWING_WAVES=5" || exit 1
done
