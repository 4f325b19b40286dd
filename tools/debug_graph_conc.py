import time, torch
dev = torch.device('cuda:0')
N = 200
xs = [torch.zeros(1 << 14, device=dev) for _ in range(2)]
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
def chain(x):
    for _ in range(N):
        x.add_(1.0)          # tiny kernel, ~few us
def big(x):
    for _ in range(N):
        torch.sin_(x)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
graphs = []
for i in range(2):
    ss[i].wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ss[i]): chain(xs[i])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=ss[i]):
        chain(xs[i])
    graphs.append(g)
def one(): graphs[0].replay()
def two_same():
    graphs[0].replay(); graphs[1].replay()
def two_streams():
    cur = torch.cuda.current_stream()
    for i in range(2):
        ss[i].wait_stream(cur)
        with torch.cuda.stream(ss[i]): graphs[i].replay()
    for i in range(2): cur.wait_stream(ss[i])
def eager_two_streams():
    cur = torch.cuda.current_stream()
    for i in range(2):
        ss[i].wait_stream(cur)
    for k in range(N):
        for i in range(2):
            with torch.cuda.stream(ss[i]): xs[i].add_(1.0)
    for i in range(2): cur.wait_stream(ss[i])
print('one graph (200 tiny kernels): %.3f ms' % timeit(one))
print('two graphs, same stream     : %.3f ms' % timeit(two_same))
print('two graphs, two streams     : %.3f ms' % timeit(two_streams))
print('eager, two streams          : %.3f ms' % timeit(eager_two_streams))
