cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tl; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O -o t -- python $GRAFT_REPO_ROOT/tools/e2e_stream_probe.py > $O/log.txt 2>&1
tail -2 $O/log.txt
DB=$(find $O -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/timeline_dump.py $DB 0 1000000 > $O/all.txt 2>&1
wc -l $O/all.txt; N=$(wc -l < $O/all.txt); sed -n "$((N/2)),$((N/2+70))p" $O/all.txt
rm -f $DB
