c Solution file for cycle_shortcut.min
c
c Optimal solution
s 50
c
c Non-zero flows (SRC DST FLOW)
f 1 4 10
c
c End of file
