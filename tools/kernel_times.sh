# per-launch-shape kernel times of a command:  bash tools/kernel_times.sh <kernel-substring> <command...>
PAT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_prof
rocprofv3 --kernel-trace --stats -d /tmp/kt_prof -o prof -- "$@" > /tmp/kt_prof.log 2>&1
DB=$(find /tmp/kt_prof -name "*results.db" | head -1)
python $ROOT/tools/rocpd_summary.py bygrid $DB /tmp/kt_bygrid.md "$PAT"
