import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
for nm,lo,hi in (("o_s",0,256),("o_pair",256,768),("o_pts",768,960),("norms",960,1024),("y",1024,1152)):
    d=np.abs(a[:,lo:hi]-b[:,lo:hi]).max(); m=np.abs(b[:,lo:hi]).max()
    print(nm, "maxabs diff %.3e"%d, "ref max %.3e"%m, "rel %.3e"%(d/m))
bad=np.abs(a[:,256:768]-b[:,256:768]); i=np.unravel_index(bad.argmax(), bad.shape); print("worst o_pair at row", i[0], "col", i[1], "(head", i[1]//64, "chan", i[1]%64, ")", a[i[0],256+i[1]], b[i[0],256+i[1]])
