# the default bench line's headline and side passes in one row (how the later trackers of the process fare)
for i in $(seq 1 ${N:-2}); do timeout 400 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-sae-pass --no-profile-pass $* 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('exact_sum_lk') or d.get('float_order_lk')
print('main', d['ms_per_step'], d['repeats']['ms_per_step'], 'other-lk', o['ms_per_step'], 'low', d['low_cpu']['one_host_thread_ms_per_step'], d['low_cpu']['two_cpus_two_threads_ms_per_step'], 'host', d['host_resident_events']['ms_per_step'], 'plain', d['one_batch_in_flight']['device_resident_ms_per_step'], d['one_batch_in_flight']['host_pageable_ms_per_step'], d['one_batch_in_flight']['host_registered_ms_per_step'])"; done
