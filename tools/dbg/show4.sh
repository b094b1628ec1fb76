#!/bin/bash
for w in "${@:2}"; do echo == $w; python3 -c "
import json;j=json.loads(open('$1/$w.json').read().strip().splitlines()[-1]);print('ms/step %.1f value %.3e'%(j['ms_per_step'],j['value']), {k:j['timing'][k] for k in ('diag_left_tasks','checked_tasks','swept_tasks','overflow_tasks')}, j['result'])"; python tools/kstats.py $1/${w}_kernel_stats.csv 4 | head -${N:-8}; done
