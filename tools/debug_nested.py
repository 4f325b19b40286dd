import sys, torch, faulthandler
faulthandler.enable()
mode = sys.argv[1]
dev = torch.device('cuda:0')
a = torch.zeros(1 << 20, device=dev)
b = torch.zeros(1 << 20, device=dev)
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def work():
    main = torch.cuda.current_stream()
    if mode == 'nested_joinmain':      # main -> s1 -> s2 ; s2 joins main only
        s1.wait_stream(main)
        with torch.cuda.stream(s1):
            a.add_(1)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                b.add_(2)
            a.add_(3)
        main.wait_stream(s1); main.wait_stream(s2)
    elif mode == 'cross':              # s1, s2 forked from main; s1 waits s2
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s2):
            b.add_(2)
        with torch.cuda.stream(s1):
            a.add_(1)
            s1.wait_stream(s2)
            a.add_(b)
        main.wait_stream(s1); main.wait_stream(s2)
    elif mode == 'flat':               # plain fork/join
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s2):
            b.add_(2)
        with torch.cuda.stream(s1):
            a.add_(1)
        main.wait_stream(s1); main.wait_stream(s2)
    elif mode == 'cross_into_main_twice':   # s2 forked twice from main at different points
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            b.add_(2)
        a.add_(1)
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            b.add_(a)
        main.wait_stream(s2)
for _ in range(2): work()
torch.cuda.synchronize()
s0.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s0): work()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s0):
    work()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print(mode, 'ok', float(a[0]), float(b[0]))
