TDR_CLIP_TOK16=1 python -m pytest tests/test_hip_i2t.py -m gpu -x -q -k "tok16x2 or token_major" 2>&1 | tail -1
for o in 0 1; do for d in 0 1; do echo "== ORDER=$o DMA=$d"; TDR_TOK16_ORDER=$o TDR_TOK16X2_DMA=$d python profiles/probe_tok16x2.py 2>&1 | sed -n 5,8p; done; done
