c Solution file for grid_5x5.min
c
c Optimal solution
s 27000
c
c Non-zero flows (SRC DST FLOW)
f 1 2 1000
f 2 3 1000
f 3 8 1000
f 8 9 1000
f 9 10 1000
f 10 15 1000
f 15 20 1000
f 20 25 1000
c
c End of file
