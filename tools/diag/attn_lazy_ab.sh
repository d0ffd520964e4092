python tools/diag/attn_small_bench.py
python - <<'PY'
import photoverse_amd.build as b
b.FLAGS = b.FLAGS + ["-DPV_ATTN_LAZY_UP=0.f"]
b.build_lib(force=True, verbose=False)
PY
echo "== eager (PV_ATTN_LAZY_UP=0)"
python tools/diag/attn_small_bench.py
python -m photoverse_amd.build --force > /dev/null
