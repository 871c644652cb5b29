#!/bin/bash
# on the GPU box: what the artefact directory's file system takes (scene.py writes 3.4 GB of .npy there)
D=${1:-/tmp}
df -hT $D | tail -2; nproc
for mode in "" "oflag=direct"; do
  rm -f $D/probe.bin
  /usr/bin/time -f "dd bs=64M count=32 $mode: %e s" dd if=/dev/zero of=$D/probe.bin bs=64M count=32 $mode 2>&1 | tail -2
done
rm -f $D/probe.bin
# four writers, one file each, buffered
t0=$(date +%s.%N); for i in 1 2 3 4; do dd if=/dev/zero of=$D/probe$i.bin bs=64M count=8 2>/dev/null & done; wait; t1=$(date +%s.%N)
echo "4 x 512 MiB buffered, parallel files: $(python3 -c "print(round(2.147/($t1-$t0),2))") GB/s"; rm -f $D/probe?.bin
python3 - <<'P'
import os, time, threading
D=os.environ.get("D","/tmp")
path=os.path.join(D,"probe_par.bin"); n=2<<30; chunk=64<<20
buf=bytes(chunk)
fd=os.open(path, os.O_WRONLY|os.O_CREAT|os.O_TRUNC, 0o644)
t0=time.time()
for off in range(0,n,chunk): os.pwrite(fd,buf,off)
t1=time.time(); print("1 thread pwrite 2 GiB buffered: %.2f GB/s"%(n/1e9/(t1-t0)))
os.ftruncate(fd,0)
def w(k,nt):
    for off in range(k*chunk,n,nt*chunk): os.pwrite(fd,buf,off)
for nt in (2,4,8):
    os.ftruncate(fd,0); t0=time.time()
    th=[threading.Thread(target=w,args=(k,nt)) for k in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]
    print("%d threads pwrite one file 2 GiB buffered: %.2f GB/s"%(nt,n/1e9/(time.time()-t0)))
os.close(fd); os.unlink(path)
P
