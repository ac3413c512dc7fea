c Solution file for grid_2x2.min
c
c Optimal solution
s 75
c
c Non-zero flows (SRC DST FLOW)
f 1 2 15
f 2 4 15
c
c End of file
