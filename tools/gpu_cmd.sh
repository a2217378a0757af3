cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PMC=1 timeout 1200 tools/collect_calibration_profiles.sh r04 1 2>&1 | tail -12
