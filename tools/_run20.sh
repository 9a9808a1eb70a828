cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r4 > gpurun_out/r4_profile_round.log 2>&1; tail -2 gpurun_out/r4_profile_round.log
bash tools/bench_records.sh r4 > gpurun_out/r4_bench_records.log 2>&1; tail -5 gpurun_out/r4_bench_records.log | cut -c1-160
python3 bench.py --mode train --batch-norm --no-accuracy --no-extra > gpurun_out/r4_bench_train_bn.json 2>/dev/null; cut -c1-200 gpurun_out/r4_bench_train_bn.json
timeout 1500 python tools/conv_clock.py --cases conv3_2:26,conv3_2:17,conv4_2:26,conv2_2:26,conv2_1:26,conv5_1:27,conv5_1:22 --out gpurun_out/r4_conv_inkernel_clock_final.json > gpurun_out/r4_conv_clock_final.txt 2>&1; tail -24 gpurun_out/r4_conv_clock_final.txt
