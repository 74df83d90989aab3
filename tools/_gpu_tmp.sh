bash tools/k1_knock.sh time "Sawyer|Pusher" 2>&1 | grep -v amdgpu
