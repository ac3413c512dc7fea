c Solution file for path_10node.min
c
c Optimal solution
s 900
c
c Non-zero flows (SRC DST FLOW)
f 1 2 20
f 2 3 20
f 3 4 20
f 4 5 20
f 5 6 20
f 6 7 20
f 7 8 20
f 8 9 20
f 9 10 20
c
c End of file
