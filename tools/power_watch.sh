# Samples the GPU's power / clocks while a command runs:  bash tools/power_watch.sh <out.txt> <command...>
OUT=$1; shift
( while true; do /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' '; echo; sleep 0.05; done ) > $OUT &
W=$!
"$@"
kill $W
