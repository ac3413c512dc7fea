c Solution file for diamond_graph.min
c
c Optimal solution
s 80
c
c Non-zero flows (SRC DST FLOW)
f 1 2 10
f 2 4 10
c
c End of file
