# the trained-scale fixture (tests/golden/ds2_cfg2_trained_summary.npz) in every precision mode: one record per mode
mkdir -p gpurun_out
cd tests
for m in ${MODES:-f16x3 bf16x3 f32 fp16}; do
  MS_PRECISION=$m timeout -k 10 300 python -c "
import cfg_checks
try:
    cfg_checks.cfg2_trained(atol=1e-3, strict_transcripts=False)
except AssertionError as e:
    print('ASSERT', e)
" > ../gpurun_out/r06_trained_$m.txt 2>&1
done
cd ..
