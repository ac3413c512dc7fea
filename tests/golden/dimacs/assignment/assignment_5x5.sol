c Solution file for assignment_5x5.min
c
c Optimal solution
s 10
c
c Non-zero flows (SRC DST FLOW)
f 1 10 1
f 2 9 1
f 3 6 1
f 4 8 1
f 5 7 1
c
c End of file