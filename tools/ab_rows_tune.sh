for t in "debug=8" "splits=13" "splits=16" "splits=19" "splits=26" "splits=32" "splits=38" "splits=64" "pre_tiles=2" "pre_tiles=4" "pre_tiles=6" "splits=19,pre_tiles=4" "splits=26,pre_tiles=4"; do
NEEDLE_BENCH_TUNE=$t timeout -k 10 120 python bench.py --rows ${ROWS:-125000} --paths none --cpu-seconds 0 --secondary-steps 0 --steps 20 --warmup 3 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d[\"config\"][\"rows_total\"], \"ms/step\", d[\"ms_per_step\"], \"scan_ms\", d[\"roofline\"][\"launch_ms\"], \"frac\", d[\"roofline\"][\"frac\"])" || exit 1
[ "$t" = "debug=8" ] && tail -1 /tmp/err.txt
done
