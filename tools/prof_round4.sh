#!/bin/bash
# Round-4 artefacts besides tools/refresh_profiles.sh (run on the GPU box): step timelines + kernel statistics of the other
# configurations, phase stamps / ablations / counters of rbvae_wgrad3x3s2_row, timings of the new kernels.  Output: gpurun_out/r04/*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for cfg in "cfg3 tools/run_cfg3.py bf16" "native tools/run_native.py bf16" "ldm tools/run_ldm.py"; do
  set -- $cfg; tag=$1; shift
  bash tools/prof_any.sh $tag "$@" > $O/${tag}_top_kernels.txt 2>&1 || exit 1
  cp gpurun_out/${tag}_kernel_stats.csv $O/${tag}_kernel_stats.csv
done
PROF_MARK="conv_first_fused_k<4, 0" bash tools/prof_step.sh native tools/run_native.py bf16 > /dev/null 2>&1; cp gpurun_out/native_step_summary.txt $O/native_step_timeline.txt
PROF_MARK="conv_first_fused_k<3, 0" bash tools/prof_step.sh cfg3 tools/run_cfg3.py bf16 > /dev/null 2>&1; cp gpurun_out/cfg3_step_summary.txt $O/cfg3_step_timeline.txt
python3 tools/time_wgrad_row.py > $O/wgrad_row_vs_gemm_times.txt 2>&1
python3 tools/time_conv_s2.py > $O/conv_s2_vs_gather_times.txt 2>&1
python3 tools/time_attn.py > $O/attention_times.txt 2>&1
python3 tools/run_ldm.py > $O/ldm_encoder_frames_per_s.txt 2>&1
bash tools/pmc_any.sh wgrad_row tools/time_wgrad_row.py 0,2 10,21 > $O/wgrad_row_pmc.txt 2>&1
for v in st stnodma stoob; do
  if [ -f symbols-from-video_amd/librbvae_hip_$v.so ]; then
    echo "== $v" >> $O/wgrad_row_stamps.txt
    RBVAE_LIB=$R/symbols-from-video_amd/librbvae_hip_$v.so python3 tools/wr_stamps.py bench2,native2 5,10,21 >> $O/wgrad_row_stamps.txt 2>&1
  fi
done
ls $O
