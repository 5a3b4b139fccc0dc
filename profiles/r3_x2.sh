for c in H L; do for t in 1 0; do echo "clip $c tok16 $t: $(TDR_CLIP_TOK16=$t python bench.py --arch i2t --clip $c --steps 20 --warmup 3 2>&1 | tail -1 | cut -c80-200)"; done; done
