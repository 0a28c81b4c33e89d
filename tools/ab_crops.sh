# interleaved A/B of executor variants over crop counts (GPU box, through gpurun): the table in hrnet_hip.py / DESIGN section 4 comes from here
for n in 12 40 60 112 217; do echo "== n=$n"; timeout 600 python tools/ab_flags.py --n $n --rounds 3 --iters 8 both: r48:block2=1 unfused:block2=0 2>&1 | tail -4; done
