import numpy as np, sys
old=np.load(sys.argv[1]); new=np.load(sys.argv[2])
same=diff=0
for k in old.files:
    if k not in new.files: print("missing in new:",k); diff+=1; continue
    a,b=old[k],new[k]
    if a.shape==b.shape and a.dtype==b.dtype and a.tobytes()==b.tobytes(): same+=1
    else: diff+=1; print("differs:",k)
print("old arrays bit-identical in the new fixture: %d, different: %d, new arrays: %d"%(same,diff,len(new.files)-len(old.files)))
