# A/B of two builds of the library on the same box: tools/ab_lib.sh <what> [n]   (expects libse_amd.old.so / libse_amd.new.so beside libse_amd.so)
set -e
cd $GRAFT_REPO_ROOT
P=speech-enhancement-by-s3prl_amd
WHAT=${1:-mhsa}; N=${2:-3}
for i in $(seq $N); do for v in old new; do cp $P/libse_amd.$v.so $P/libse_amd.so; echo -n "$v: "; timeout -k 5 120 python3 tools/bench_kernels.py $WHAT 2>&1 | grep -v "^$" | cut -c1-130 | tail -${3:-1}; done; done
cp $P/libse_amd.new.so $P/libse_amd.so
