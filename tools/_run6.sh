mkdir -p gpurun_out/r05_v61
# the quality gate once on HEAD (the recommended recipe) ...
python examples/selfplay_train.py --games 65536 --rounds 16 --slots 2048 --scale-games 96 --continuous --classic-rounds 3 --pipeline-rounds > gpurun_out/r05_v61/quality_cont_pipe.txt 2>&1; grep -v amdgpu gpurun_out/r05_v61/quality_cont_pipe.txt | tail -5 | cut -c1-300
# ... and the two-rank rehearsal of the training CLI's multi-rank loops on the one GPU (gloo): classic rounds, sequential windows with held-back games (the hold decision is
# collective), pipelined windows
for mode in "--continuous --classic-rounds 1 --min-window-games 1500" "--continuous --classic-rounds 1 --pipeline-rounds"; do
  echo "== 2 ranks: $mode"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 examples/selfplay_train.py --dist-backend gloo --games 2048 --rounds 5 --slots 512 --scale-games 96 --arena 256 $mode 2>&1 | grep -v "amdgpu\|socket.cpp\|Gloo\|^\*\*\*\|^W1005\|OMP_NUM" | tail -6 | cut -c1-300
done > gpurun_out/r05_v61/selfplay_train_2ranks_gloo.txt 2>&1; cat gpurun_out/r05_v61/selfplay_train_2ranks_gloo.txt
