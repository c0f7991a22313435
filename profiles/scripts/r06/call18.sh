#!/bin/bash
# r06 call 18: k_attn_fa with a tile's V fragments requested at the top of the tile (second library build, -DFA_VPRE=8): parity, kernel durations
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
T0=$(date +%s)
ALT=$R/vita_amd/lib/libvita_hip_vpre8.so
(cd $R && VITA_AMD_LIB=$ALT timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "variants or flash_form" > $O/call18_pytest.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call18_status.txt)
tail -2 $O/call18_pytest.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --steps 8 --warmup 2"
for arm in base vpre8; do
  if [ $arm = base ]; then unset VITA_AMD_LIB; else export VITA_AMD_LIB=$ALT; fi
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- $B > $O/c18_tp1_$arm.json 2>> $O/c18.err)
  python3 $R/profiles/summarize.py "$(find /tmp/kt -name '*.db' | head -1)" 'k_attn_fa' > $O/c18_kstats_tp1_$arm.txt
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- $B --frames 8 > $O/c18_frames8_$arm.json 2>> $O/c18.err)
  python3 $R/profiles/summarize.py "$(find /tmp/kt -name '*.db' | head -1)" 'k_attn_fa' > $O/c18_kstats_frames8_$arm.txt
done
unset VITA_AMD_LIB
for f in $O/c18_kstats_*.txt; do echo "== $(basename $f)"; grep -v "^#" $f | head -3 | cut -c1-170; done
echo "total $(( $(date +%s) - T0 )) s"
