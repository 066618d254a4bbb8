#!/bin/bash
# Shader clock / power / temperature of the box's GPU sampled every ~50 ms while a command runs.   Usage: tools/clock_sample.sh <out.txt> <command ...>
# Reads the amdgpu sysfs nodes (pp_dpm_sclk: the line marked '*' is the current level; hwmon power1_average / power1_input in microwatts) and, when they are absent,
# falls back to one `rocm-smi --showclocks --showpower` per 0.5 s.  The command's stdout goes to <out.txt>.cmd.
OUT=$1; shift
DEV=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
HW=$(ls -d $DEV/hwmon/hwmon* 2>/dev/null | head -1)
"$@" > $OUT.cmd 2> $OUT.err &
PID=$!
: > $OUT
echo "# t_ms sclk_mhz power_w (sysfs $DEV)" >> $OUT
T0=$(date +%s%N)
while kill -0 $PID 2>/dev/null; do
  NOW=$(( ($(date +%s%N) - T0) / 1000000 ))
  if [ -r "$DEV/pp_dpm_sclk" ]; then
    S=$(grep '\*' $DEV/pp_dpm_sclk | sed 's/.*: *\([0-9]*\)Mhz.*/\1/' | tr '\n' ' ')
    P=""
    for f in $HW/power1_average $HW/power1_input; do [ -r $f ] && P=$(( $(cat $f) / 1000000 )) && break; done
    F=""
    [ -r $HW/freq1_input ] && F=$(( $(cat $HW/freq1_input) / 1000000 ))
    echo "$NOW $S $P $F" >> $OUT
    sleep 0.05
  else
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' | sed "s/^/$NOW /" >> $OUT; echo >> $OUT
    sleep 0.5
  fi
done
wait $PID
echo "# command rc $?" >> $OUT
