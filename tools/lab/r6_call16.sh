#!/bin/bash
cd $GRAFT_REPO_ROOT/vae-gslm_amd
O=$GRAFT_REPO_ROOT/gpurun_out/r6c16; mkdir -p $O
( time timeout 600 python -m scripts.train -c configs/train/speech/vae-gslm.yaml --synthetic --max_steps 30 ) > $O/train.txt 2>&1; echo "rc=$?" >> $O/train.txt
tail -12 $O/train.txt
( time timeout 600 python -m scripts.infer -c configs/infer/speech/vae-gslm.yaml --synthetic --batch 8 ) > $O/infer.txt 2>&1; echo "rc=$?" >> $O/infer.txt
tail -6 $O/infer.txt
