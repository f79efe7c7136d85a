set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05/alt
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
MTVAF_UNPAD=0 timeout -k 10 560 python -m pytest tests -m gpu -q > $O/padded.log 2>&1; echo "MTVAF_UNPAD=0 rc=$?"; tail -1 $O/padded.log
MTVAF_F32_SPLIT=0 timeout -k 10 560 python -m pytest tests -m gpu -q > $O/pipe.log 2>&1; echo "MTVAF_F32_SPLIT=0 rc=$?"; tail -1 $O/pipe.log
