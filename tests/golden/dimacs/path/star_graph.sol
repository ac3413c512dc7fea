c Solution file for star_graph.min
c
c Optimal solution
s 90
c
c Non-zero flows (SRC DST FLOW)
f 2 1 5
f 3 1 5
f 4 1 5
f 1 5 15
c
c End of file
