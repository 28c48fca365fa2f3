cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "$1"; env $1 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   ', round(d['value'],1), round(d['ms_per_step'],3), d['param_checksum'])"; tail -1 gpurun_out/err.txt | grep -v timed; }
P1="depth,pose_encoder:encoder.conv1,beam_encoder_pose:encoder.conv1"
P2="depth,pose_encoder:encoder.conv1,beam_encoder_pose:encoder.conv1,pose_encoder:encoder.layer1,beam_encoder_pose:encoder.layer1"
P3="depth,pose_encoder:encoder.conv1,beam_encoder_pose:encoder.conv1,encoder:encoder.conv1,beam_encoder:encoder.conv1"
for i in 1 2; do
run FD_SIDE_WGRAD=depth
run FD_SIDE_WGRAD=$P1
run FD_SIDE_WGRAD=$P2
run FD_SIDE_WGRAD=$P3
done
