# test/sql/gh-2.sql of the reference: a scan of an empty index returns no rows
seqscan off
create_table t
create_index t t_val_idx l2 dims=3,m=3
select t <-> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <-> array[3,3,3];
drop_table t
