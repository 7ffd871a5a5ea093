#!/bin/bash
# The op_sel loop of e8_quad_256x256 with the next quad's words moved inside the blocks (product) against ten moves + s_nop between two quads
# (since the result: the product is the old form, build the variant with DG_VARIANT=qs DG_VARIANT_FLAGS=-DDG_QUAD_SPREAD=1 and swap the tags): dense_ue8m0 / dense_sm100 / contiguous_ue8m0 alternating on one box
run() { python bench.py --workload $1 --steps 200 --warmup 50 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$2 $1', round(d['roofline']['kernel_us'], 2), 'us', d['config'].get('kernel'))"; }
for i in 1 2 3; do
  for w in dense_ue8m0 contiguous_ue8m0; do
    run $w product
    DG_VARIANT=noqs run $w noqs
  done
done
