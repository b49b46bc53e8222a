# round 5, call U: is the batch-1 latency regime (4.0 vs 4.95 ms, box by box) the NUMA distance between the process and its GPU?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_u}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_latency_numa.txt
: > $OUT
for n in /sys/devices/system/node/node*; do echo "$(basename $n) cpus $(cat $n/cpulist)" >> $OUT; done
python - >> $OUT 2>&1 <<'P'
import os, glob, torch
p = torch.cuda.get_device_properties(0)
pci = "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
print("gpu pci", pci)
for d in glob.glob("/sys/bus/pci/devices/%s*" % pci):
    print(d, "numa_node", open(d + "/numa_node").read().strip(), "local_cpulist", open(d + "/local_cpulist").read().strip())
print("affinity of this process:", len(os.sched_getaffinity(0)), "cpus", sorted(os.sched_getaffinity(0))[:4], "...")
P
which taskset numactl >> $OUT 2>&1
BENCH="python bench.py --config c2 --batch 1 --streams 1 --steps 100 --warmup 40 --profile-steps 0 --no-cpu-baseline --no-f32-variant --no-other-configs"
run() { export LABEL="$1"; shift; timeout 120 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('latency', os.environ['LABEL'], d['ms_per_step'], 'host', d['config'].get('host_enqueue_ms_per_step'))" >> $OUT; }
for rep in 1 2; do
  run "unbound" $BENCH
  for n in /sys/devices/system/node/node*; do
    run "taskset $(basename $n)" taskset -c $(cat $n/cpulist) $BENCH
  done
done
cat $OUT
