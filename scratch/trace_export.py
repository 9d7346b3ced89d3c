import csv, sys
w = open(sys.argv[2], "w")
for r in csv.DictReader(open(sys.argv[1])):
    w.write("%s\t%s\t%s\t%s\t%s\n" % (r["Queue_Id"], r["Stream_Id"], r["Start_Timestamp"], r["End_Timestamp"], r["Kernel_Name"][:120]))
