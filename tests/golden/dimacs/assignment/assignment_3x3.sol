c Solution file for assignment_3x3.min
c
c Optimal solution
s 5
c
c Non-zero flows (SRC DST FLOW)
f 1 4 1
f 2 5 1
f 3 6 1
c
c End of file
