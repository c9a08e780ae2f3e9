set -e
O=gpurun_out/r4h; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py --no-cpu-baseline > /dev/null 2>&1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/cfg3.json 2> $O/cfg3.err; python -c "import json;d=json.loads(open('$O/cfg3.json').read().strip().splitlines()[-1]);print('cfg3',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_bn']['frac'])"
for w in cfg2 cfg4; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 > $O/$w.json 2> $O/$w.err; python -c "import json;d=json.loads(open('$O/$w.json').read().strip().splitlines()[-1]);print('$w',d['value'],d['ms_per_step'])"; done
timeout -k 10 300 python bench.py --batch 64 --no-cpu-baseline --steps 10 --warmup 3 > $O/cfg3_b64.json 2> $O/b64.err; python -c "import json;d=json.loads(open('$O/cfg3_b64.json').read().strip().splitlines()[-1]);print('b64',d['value'],d['ms_per_step'])"
for w in cfg2 cfg4; do timeout -k 10 300 python bench.py --workload $w --eval-amp --no-cpu-baseline --steps 10 --warmup 3 > $O/${w}_allbf16.json 2> $O/${w}a.err; python -c "import json;d=json.loads(open('$O/${w}_allbf16.json').read().strip().splitlines()[-1]);print('$w allbf16',d['value'],d['ms_per_step'])"; done
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/$O/prof.log 2>&1; cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof; ls -la $O
