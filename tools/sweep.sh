run() { echo "== $*"; env "$@" python tools/tlrun.py 2>&1 | grep ms/step; }
for i in 1 2; do
run A=1
run RAU_FWD_CAP_HOPS=2
run RAU_FWD_CAP_HOPS=4
run RAU_FWD_CAP_HOPS=6
done
