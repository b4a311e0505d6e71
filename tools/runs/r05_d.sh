mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "6144 512 512 asm" "6144 512 512 hip" "11520 1536 512 asm" "11520 1536 512 hip" "59648 1536 512 asm" "59648 1536 512 hip"; do
  tag=$(echo $spec | tr ' ' '_')
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o x -- python $R/tools/one_gemm.py $spec > /tmp/o.log 2>&1
  DB=$(find /tmp/pp -name "*.db" | head -1)
  python $R/tools/prof_summarize.py stats "$DB" /tmp/s.txt "$spec" | sed -n 4,6p > $R/gpurun_out/one_$tag.txt
  echo "== $spec"; cat $R/gpurun_out/one_$tag.txt
done
