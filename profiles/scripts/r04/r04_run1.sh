#!/bin/bash
# r04, GPU call 1: parity of the specialised streaming GEMM (ps_cfg = 2), interleaved A/B against ps_cfg = 1 (uniform and skewed
# routing), ablations of the new kernel.  Run through gpurun from the repo root; output under gpurun_out/r04_run1/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run1; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/env.txt 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm_ps" > $O/pytest_gemm_ps.log 2>&1; echo "pytest gemm_ps rc=$?" | tee -a $O/env.txt
timeout 400 python profiles/bench_moe_gemm.py --ab 1,2 --rounds 3 > $O/ab_uniform.log 2>&1; echo "ab uniform rc=$?" | tee -a $O/env.txt
timeout 400 python profiles/bench_moe_gemm.py --ab 1,2 --rounds 3 --skew > $O/ab_skew.log 2>&1; echo "ab skew rc=$?" | tee -a $O/env.txt
for n in 1 2 3 4 7; do
  VITA_AMD_LIB=$R/build/abl/libvita_hip_sp_$n.so timeout 300 python profiles/bench_moe_gemm.py --nocheck --ab 2 --rounds 2 > $O/abl_sp_$n.log 2>&1; echo "ablate $n rc=$?" | tee -a $O/env.txt
done
grep -h "round\|cfg=" $O/ab_uniform.log $O/ab_skew.log | cut -c1-160
for n in 1 2 3 4 7; do echo "ABLATE $n"; grep -h "round" $O/abl_sp_$n.log | cut -c1-160; done
