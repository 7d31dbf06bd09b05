# test/sql/knn.sql of the reference, statement by statement (explain lines: see "cost")
seqscan off
create_table t
insert t {0,1,2}
insert t {1,2,3}
insert t {1,1,1}
insert t NULL
create_index t t_val_idx l2 dims=3,m=3
insert t {1,2,4}
cost t t_val_idx
select t <-> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <-> array[3,3,3];
count t ; SELECT COUNT(*) FROM t;
create_index t t_val_idx1 cos dims=3,m=3
select t <=> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <=> array[3,3,3];
create_index t t_val_idx2 manhattan dims=3,m=3
select t <~> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <~> array[3,3,3];
seqscan on
select t <-> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <-> array[3,3,3];
select t <=> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <=> array[3,3,3];
select t <~> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <~> array[3,3,3];
delete_all t
vacuum t
insert t {0,1,2}
insert t {1,2,3}
insert t {1,1,1}
insert t NULL
insert t {1,2,4}
seqscan off
select t <-> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <-> array[3,3,3];
select t <=> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <=> array[3,3,3];
select t <~> {3,3,3} val 0 ; SELECT * FROM t ORDER BY val <~> array[3,3,3];
drop_table t
