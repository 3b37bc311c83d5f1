run() { echo "== $*"; env "$@" python tools/tlrun.py 2>&1 | grep ms/step; }
for i in 1 2; do
run RAU_CONV_SAMPLE=12
run RAU_CONV_SAMPLE=8
run RAU_CONV_SAMPLE=14
run RAU_CONV_SAMPLE=15
run RAU_CONV_SAMPLE=13
done
