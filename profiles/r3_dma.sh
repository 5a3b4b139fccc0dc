for c in 0 1; do for r in 1 4 8 16; do echo "== CFG=$c ROT=$r"; TDR_TOK16_ROT=$r TDR_TOK16_CFG=$c python profiles/probe_tok16.py 2>&1 | sed -n 2,5p; done; done
