#!/bin/bash
# The sweep behind profiles/round4_skip_ablation.log: the step with the launches of one kernel class dropped (run on the GPU box after scripts/build_skip_ablation.sh)
cd $GRAFT_REPO_ROOT
export FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_skip.so
L=gpurun_out/r4_skip_ablation2.log
: > $L
run() { FD_SKIP="$2" timeout 120 python scripts/step_time.py "$1" >> $L 2>/dev/null || echo "$1 FAILED" >> $L; }
P="k_input_normalize"      # garbage into every encoder: the data effect (power / clock) alone
BN="norm.hip"
FIN="k_wgrad_finish,k_splitk_finish,k_reduce_slabs,k_channel_sum,k_wino2d_finish"
POOL="k_maxpool,k_upcat,k_slice,k_up2_,k_act_bwd,k_axpby,k_spatial,k_reflect,k_zero_border"
LOSS="photometric_ms.hip,smooth.hip,geometry.hip"
CFAST="conv_fast.hip:574,conv_fast.hip:636"
WFAST="conv_fast.hip:747"
run baseline ""
run P_polluted "$P"
run P+bn_fwd "$P,norm.hip:533,norm.hip:542,norm.hip:545,norm.hip:566"
run P+bn_bwd "$P,norm.hip:604,norm.hip:612,norm.hip:616"
run P+finishers "$P,$FIN"
run P+pool "$P,$POOL"
run P+loss "$P,$LOSS"
run P+relayout "$P,k_relayout_batch"
run relayout "k_relayout_batch"
FD_LATE_RELAYOUT=0 run late0_baseline ""
FD_LATE_RELAYOUT=0 run late0_relayout "k_relayout_batch"
run P+all_non_mfma "$P,$BN,$FIN,$POOL,$LOSS,k_relayout_batch"
run P+wgrad_wino "$P,k_wgrad_wino"
run P+conv_wino2d "$P,k_conv_wino2d"
run P+conv_wino2p "$P,k_conv_wino2p"
run P+conv_fast "$P,$CFAST"
run P+wgrad_fast "$P,$WFAST"
run P+stems "$P,k_conv7s2_stem,k_wgrad_stem"
run P+narrow_n16_gather "$P,k_wgrad_narrow,k_conv3x3_n16,conv.hip:640,conv.hip:687,conv_c1.hip"
run P+adam "$P,k_adam_dev"
run P_again "$P"
run baseline_again ""
cat $L
