for cfg in "" "--config 2 --channels 8"; do
echo "== $cfg"
QI_TUNE=1 QI_NATIVE_VERBOSE=1 python bench.py $cfg --cpu-seconds 0 --steps 1 --warmup 1 --settle-ms 0 --wrappers 0 2>&1 | grep "zoom launch" | sort | uniq -c
done
