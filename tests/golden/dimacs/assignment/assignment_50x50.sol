c Solution file for assignment_50x50.min
c
c Optimal solution
s 50
c
c Non-zero flows (SRC DST FLOW)
f 1 73 1
f 2 62 1
f 3 80 1
f 4 52 1
f 5 85 1
f 6 96 1
f 7 98 1
f 8 76 1
f 9 56 1
f 10 54 1
f 11 66 1
f 12 55 1
f 13 64 1
f 14 60 1
f 15 89 1
f 16 65 1
f 17 71 1
f 18 84 1
f 19 82 1
f 20 86 1
f 21 91 1
f 22 51 1
f 23 90 1
f 24 93 1
f 25 69 1
f 26 97 1
f 27 70 1
f 28 53 1
f 29 61 1
f 30 75 1
f 31 72 1
f 32 100 1
f 33 74 1
f 34 83 1
f 35 81 1
f 36 94 1
f 37 67 1
f 38 58 1
f 39 79 1
f 40 95 1
f 41 77 1
f 42 99 1
f 43 87 1
f 44 63 1
f 45 57 1
f 46 88 1
f 47 92 1
f 48 59 1
f 49 78 1
f 50 68 1
c
c End of file
