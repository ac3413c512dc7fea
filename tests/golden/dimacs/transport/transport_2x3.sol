c Solution file for transport_2x3.min
c
c Optimal solution
s 85
c
c Non-zero flows (SRC DST FLOW)
f 1 3 15
f 1 5 5
f 2 4 20
f 2 5 10
c
c End of file
