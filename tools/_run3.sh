cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "bucket" 2>&1 | tail -25 > $O/r6c_tests.txt
python3 -m pytest tests/test_models_gpu.py -x -q -m gpu -k "padded or 129" 2>&1 | tail -8 >> $O/r6c_tests.txt
python3 bench.py --record dirichlet_fit > $O/r6c_fit.json 2> $O/r6c_fit.err
python3 bench.py --gpus 8 --share-device --dist-backend gloo --mode train --batch-norm --batch 4 --steps 5 --warmup 2 --min-seconds 0 > $O/r6_dp8_gloo_train_bn.json 2> $O/r6c_dp8.err
cat $O/r6c_tests.txt; cut -c1-1200 $O/r6c_fit.json; tail -3 $O/r6c_fit.err; cut -c1-2500 $O/r6_dp8_gloo_train_bn.json; tail -5 $O/r6c_dp8.err
