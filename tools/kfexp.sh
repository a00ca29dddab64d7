# timing experiments for the candidate filter (PD_KF_DEBUG: 1 = stop before the column pass)
export TMPDIR=/tmp
for v in "PD_KF_DEBUG=0" "PD_KF_DEBUG=1" "PD_KF_DEBUG=2"; do
  echo "== $v"
  rm -rf /tmp/kt; env $v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python bench.py --steps 40 --warmup 4 --contexts 1 --no-cpu > /tmp/kt.log 2>&1
  cat /tmp/kt/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-90
done
