OUT=gpurun_out/r06f; mkdir -p $OUT
python tools/chain_ab.py 8 96 12 7 > $OUT/chain_ab_8x96.txt 2>&1
python tools/chain_ab.py 4 96 12 7 > $OUT/chain_ab_4x96.txt 2>&1
python tools/chain_ab.py 16 96 12 7 > $OUT/chain_ab_16x96.txt 2>&1
python tools/chain_ab.py 4 160 12 7 > $OUT/chain_ab_4x160.txt 2>&1
cat $OUT/chain_ab_*.txt | grep -v "^/opt"
