cd "$(dirname "$0")/../.."
for v in NOFOLDST NOFOLDLD NOSTAGE; do
scripts/dev/build_variant.sh conv_bwd_x6 /tmp/lib_pv.so -DX6B_PROF -DX6B_DBG_$v 2>/dev/null && echo "== $v" && MFVI_LIB_PATH=/tmp/lib_pv.so timeout -k 10 200 python3 scripts/dev/bwdx6_prof.py 2>/dev/null | grep -E "us|s\.fold|s\.stage|m\.wait B1|m\.rows"
done
