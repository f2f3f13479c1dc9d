python tools/split_check.py 2>&1 | tail -2
python tools/split_net_check.py w48 64 2>&1 | grep -A6 "variant -1\|output"
python tools/split_net_check.py det 16 2>&1 | grep -A6 "variant -1" 
python tools/split_net_check.py roi 32768 2>&1 | grep -A4 "variant\|cls:\|reg:" 
