export CRT_WATCHDOG_SECONDS=5
timeout -k 10 150 python -u -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 100 python bench.py --steps 3 --cpu-spp 0 --no-roofline-counters 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('bench ms', d['ms_per_step'], d['value'])" || exit 1
CRT_COUNTED_BY_MASK=1 timeout -k 10 100 python tools/sched_probe.py 8 2>&1 | grep -E "^box|^triangle|^transform|^path|^exchange|regrouping|lanes at a vote|queue fill|kernel ms"
