# cache policy of the row kernels' bulk stores (rowwalk.h FV_BUF_STORE_AUX): 0 default, 2 nt, 16 sc1, 17 sc0 sc1
mkdir -p gpurun_out/r04l
for aux in 0 16 2 17 0; do
  FASTVIM_EXTRA_FLAGS="-DFV_BUF_STORE_AUX=$aux" python -m fastvim_amd.build --force > gpurun_out/r04l/build_$aux.log 2>&1 || tail -3 gpurun_out/r04l/build_$aux.log
  for i in 1 2; do echo -n "aux $aux: "; python bench.py --steps 20 --warmup 5 --no-other-configs --no-scan-op --no-cpu-baseline --no-kernels 2>/dev/null | grep -o '"ms_per_step": [^,]*\|"final_loss_hex": "[^"]*"' | tr '\n' ' '; echo; done
done | tee gpurun_out/r04l/store_policy.log
