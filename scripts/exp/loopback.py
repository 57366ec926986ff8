import socket, threading, time
def srv(ls, n, sz):
    c,_=ls.accept(); got=0; buf=bytearray(1<<20)
    while got<n*sz:
        k=c.recv_into(buf); 
        if not k: break
        got+=k
ls=socket.socket(); ls.bind(("127.0.0.1",0)); ls.listen(1); port=ls.getsockname()[1]
n,sz=2000,475000
t=threading.Thread(target=srv,args=(ls,n,sz)); t.start()
c=socket.socket(); c.connect(("127.0.0.1",port)); c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
b=bytes(sz); t0=time.time()
for i in range(n): c.sendall(b)
c.close(); t.join(); dt=time.time()-t0
print("loopback single stream: %.2f GB/s" % (n*sz/dt/1e9))
