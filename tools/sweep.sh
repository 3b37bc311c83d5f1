run() { echo "== $*"; env "$@" python tools/tlrun.py 2>&1 | grep ms/step; }
for i in 1 2; do
run A=1
run RAU_SKINNY_WGS=64
run RAU_SKINNY_WGS=96
run RAU_SKINNY_WGS=128
run RAU_SKINNY_WGS=200
done
