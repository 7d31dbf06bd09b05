# test/sql/gh-3.sql of the reference: TRUNCATE rebuilds the index
seqscan off
create_table t serial
create_index t t_val_idx l2 dims=3,m=3
insert t {0,1,2}
insert t {1,2,3}
insert t {1,1,1}
truncate t
insert t {4,5,6}
insert t {1,2,3}
insert t {7,8,9}
select t <-> {3,3,3} ctid,id 0 ; SELECT ctid, id from t order by val <-> ARRAY[3,3,3];
drop_table t
