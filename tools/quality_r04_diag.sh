run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 --games 65536 --max-plies 400 --slots 2048 --scale-games 96 --verbose "$@" 2>&1 | grep -v amdgpu.ids | grep "window\|round\|after\|games,"; }
run --rounds 16 --continuous --classic-rounds 3
run --rounds 16 --continuous --classic-rounds 3 --seed 2
run --rounds 16 --continuous --classic-rounds 3 --pipeline-rounds
run --rounds 16 --continuous --classic-rounds 3 --pipeline-rounds --seed 2
run --rounds 29 --continuous --classic-rounds 3 --pipeline-rounds --window-steps 42
run --rounds 17 --continuous --min-window-games 65536 --seed 2
run --rounds 16 --continuous --burn-in-windows 3 --scale-warmup 24
run --rounds 16 --continuous --classic-rounds 2 --pipeline-rounds --seed 3
