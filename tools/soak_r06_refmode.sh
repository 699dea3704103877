s() { python tools/soak.py "$@" 2>&1 | grep -i soak; }
echo "== reference mode on the lane passes, round-6 final build (explicitly fused per-point block; lane pairs below 32769 instances): QuatMpc chunks of 20480 / 32768 / 65536"
s --mode 1 --instances 204800 --chunk 20480 --horizon 10; s --mode 1 --instances 262144 --chunk 32768 --horizon 10; s --mode 1 --instances 131072 --chunk 65536 --horizon 10
s --mode 1 --instances 131072 --chunk 16384 --horizon 20; s --mode 1 --instances 131072 --chunk 32768 --horizon 20; s --mode 1 --instances 131072 --chunk 65536 --horizon 20; s --mode 1 --instances 65536 --chunk 32768 --horizon 24
echo "== ... ConvexMpc N=20 / N=10, 8-point model N=16 (plain forms)"
s --model convex --mode 1 --instances 131072 --chunk 65536 --horizon 20; s --model convex --mode 1 --instances 131072 --chunk 32768 --horizon 10; s --model biped8 --mode 1 --instances 65536 --chunk 65536 --horizon 16
echo "== closed loop in the reference's solver mode, 65536 robots on the lane passes"
s --closed-loop --mode 1 --robots 65536 --ticks 100 --feed-ang-vel
