c Solution file for path_5node.min
c
c Optimal solution
s 100
c
c Non-zero flows (SRC DST FLOW)
f 1 2 10
f 2 3 10
f 3 4 10
f 4 5 10
c
c End of file
