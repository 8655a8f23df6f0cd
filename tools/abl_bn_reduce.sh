# What the step would gain if the BatchNorm-backward reduce pass cost nothing (first half): a timing-only
# build whose reduce kernels read nothing and report zero sums (tools/build_variant.sh bnabl4 az_bn3d.hip -DBN_ABL=4), so the apply
# pass writes dx = k0 dz -- finite gradients of the usual size; both arms with AZ_PRESPLIT=0 (the pass also produces the bound of
# the pre-split form).  The first attempt skipped the launch altogether (-DBN_ABL=1 / 3): stale partial sums, the weights blew up
# after one step (loss 22.549 in every run) and the backward pass multiplied degenerate numbers at a higher clock: -10.9 ms, not
# a measurement of this pass.  Same box, interleaved.
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --eager-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'loss', d['loss'])"; }
for r in 1 2; do
  echo "shipped                        $(run)"
  echo "AZ_PRESPLIT=0                  $(AZ_PRESPLIT=0 run)"
  echo "AZ_PRESPLIT=0, free reduce     $(AZ_PRESPLIT=0 AZ_LIB_PATH=$PWD/activezero_amd/lib/variants/libazhip_bnabl4.so run)"
done
