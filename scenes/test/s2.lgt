// lights count
2

// Position                Total flux      RGB coefficients (not normalized)
-0.2000  1.5500  2.9900    5000.0000       1.0000 0.9800 0.9500
 0.0765  2.8580  0.5513    5000.0000       0.3333 0.3333 0.3333

