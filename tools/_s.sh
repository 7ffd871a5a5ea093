cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in 1 2; do
for c in e8_quad_256x256 e8_octo_256x256; do
  timeout 120 python bench.py --workload dense_ue8m0 --config $c --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>&1 | tail -1 | python -c "import json,sys; p=json.loads(sys.stdin.read()); print('$c', round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], p['calc_diff_vs_reference_expr'])"
done; done
